"""CPU oracle for the LLaVA-1.6 (LlavaNext + Mistral) reward-scoring path -- TEST INFRASTRUCTURE ONLY.

Restates, in plain torch fp32, what the reference's `custom_forward` does for model_type == 'llava'
(llava_reward/models/rw_model_general_preference.py:372-375 then :407-448): it calls
`LlavaNextForConditionalGeneration.forward(**inputs_batch, output_hidden_states=True)` and applies the value head to
`hidden_states[-1]`.  The backbone is THIRD PARTY, not vendored in /root/reference: transformers (pinned 4.50.0 in
requirements.txt:9; the container has 5.15.0) -- modeling_llava_next.py (get_image_features, pack_image_features,
unpad_image, masked_scatter of image rows), modeling_clip.py (tower, hidden_states[-2], CLS dropped) and
modeling_mistral.py (pre-norm decoder, GQA, RoPE theta 1e6, SwiGLU; position_ids = arange(S) because the
reference passes none; causal + padding mask).
Parity pin: tests/golden/ref_llava_*.json, produced by tests/golden/make_goldens.py from the reference's
own custom_forward running on the container's transformers (version skew documented in SURVEY.md §8c).
"""
from __future__ import annotations

import math
from typing import Callable, Dict, Optional

import torch
import torch.nn.functional as F

from . import phi3v_reward_oracle as po

Ident = po.Ident
CLIP_PREFIX = "vision_tower.vision_model."


def image_rows(W, cfg, pixel_values, image_sizes, opr=Ident):
    """modeling_llava_next.py get_image_features + pack_image_features: per image, rows = [base crop tokens;
    un-padded hi-res grid with image_newline appended to every row].  Returns (rows [sum V, D], counts)."""
    from llava_reward_amd.synth import llava_geometry
    g = cfg.clip.grid
    geo = [llava_geometry(int(h), int(w), cfg.pinpoints, cfg.clip.image, g) for h, w in image_sizes]
    crops = torch.cat([pixel_values[b, : 1 + geo[b][0] * geo[b][1]] for b in range(pixel_values.shape[0])], dim=0)
    feats = po.clip_tower(W, crops, cfg.clip, opr, prefix=CLIP_PREFIX)                      # [NC, g*g, Hc]
    x = po._lin(feats, W["multi_modal_projector.linear_1.weight"], W["multi_modal_projector.linear_1.bias"], opr)
    x = F.gelu(x)
    x = po._lin(x, W["multi_modal_projector.linear_2.weight"], W["multi_modal_projector.linear_2.bias"], opr)
    nl = W["image_newline"]
    out, counts, c0 = [], [], 0
    for gh, gw, r0, r1, cc0, cc1, n in geo:
        base = x[c0]
        hi = x[c0 + 1: c0 + 1 + gh * gw].view(gh, gw, g, g, -1).permute(0, 2, 1, 3, 4).reshape(gh * g, gw * g, -1)
        hi = hi[r0:r1, cc0:cc1]
        hi = torch.cat([hi, nl.expand(hi.shape[0], 1, -1)], dim=1).reshape(-1, hi.shape[-1])
        rows = torch.cat([base, hi], dim=0)
        assert rows.shape[0] == n
        out.append(rows)
        counts.append(n)
        c0 += 1 + gh * gw
    return torch.cat(out, dim=0), counts


def rope_cos_sin(position_ids, head_dim, theta):
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.int64).float() / head_dim))
    fr = position_ids[:, :, None].float() * inv[None, None, :]
    emb = torch.cat((fr, fr), dim=-1)
    return emb.cos(), emb.sin()


def decoder_layer(W, l, x, mask4d, cos, sin, cfg, opr=Ident):
    """modeling_mistral.py MistralDecoderLayer (eager attention, repeat_kv for GQA)."""
    p = f"language_model.model.layers.{l}."
    B, S, D = x.shape
    H, KV, hd = cfg.heads, cfg.kv_heads, cfg.head_dim
    h = po.rms_norm(x, W[p + "input_layernorm.weight"], cfg.rms_eps)
    q = po.proj(W, p + "self_attn.q_proj", h, None, opr).view(B, S, H, hd).transpose(1, 2)
    k = po.proj(W, p + "self_attn.k_proj", h, None, opr).view(B, S, KV, hd).transpose(1, 2)
    v = po.proj(W, p + "self_attn.v_proj", h, None, opr).view(B, S, KV, hd).transpose(1, 2)
    c, s = cos[:, None], sin[:, None]
    q = q * c + po.rotate_half(q) * s
    k = k * c + po.rotate_half(k) * s
    k = k.repeat_interleave(H // KV, dim=1)
    v = v.repeat_interleave(H // KV, dim=1)
    att = torch.matmul(opr(q), opr(k).transpose(2, 3)) / math.sqrt(hd) + mask4d
    att = torch.softmax(att, dim=-1, dtype=torch.float32)
    o = torch.matmul(opr(att), opr(v)).transpose(1, 2).reshape(B, S, H * hd)
    x = x + po.proj(W, p + "self_attn.o_proj", o, None, opr)
    h = po.rms_norm(x, W[p + "post_attention_layernorm.weight"], cfg.rms_eps)
    gate = po.proj(W, p + "mlp.gate_proj", h, None, opr)
    up = po.proj(W, p + "mlp.up_proj", h, None, opr)
    return x + po.proj(W, p + "mlp.down_proj", F.silu(gate) * up, None, opr)


@torch.no_grad()
def custom_forward(W: Dict[str, torch.Tensor], cfg, input_ids, attention_mask, pixel_values, image_sizes,
                   training: bool = False, opr: Callable = Ident, taps: Optional[dict] = None,
                   mean_hidden_state: bool = False) -> torch.Tensor:
    input_ids = torch.as_tensor(input_ids)
    attention_mask = torch.as_tensor(attention_mask)
    pixel_values = torch.as_tensor(pixel_values, dtype=torch.float32)
    B, S = input_ids.shape
    x = W["language_model.model.embed_tokens.weight"][input_ids]
    rows, counts = image_rows(W, cfg, pixel_values, torch.as_tensor(image_sizes).tolist(), opr)
    slot = input_ids == cfg.image_token_id
    assert int(slot.sum()) == rows.shape[0], "Image features and image tokens do not match"
    x = x.clone()
    x[slot] = rows                                          # masked_scatter, row-major order
    if taps is not None:
        taps["image_rows"], taps["embeds"] = rows, x.clone()
    pos = torch.arange(S)[None].expand(B, S)                # position_ids=None -> arange (modeling_mistral.py:359-362)
    cos, sin = rope_cos_sin(pos, cfg.head_dim, cfg.rope_theta)
    mask4d = po.causal_padding_mask(attention_mask)
    for l in range(cfg.layers):
        x = decoder_layer(W, l, x, mask4d, cos, sin, cfg, opr)
        if taps is not None:
            taps[f"layer{l}"] = x.clone()
    h = po.rms_norm(x, W["language_model.model.norm.weight"], cfg.rms_eps)      # hidden_states[-1]
    if mean_hidden_state:                                    # rw_model:398-406
        return F.linear(po.mean_pool(h, attention_mask), W["value_head.weight"])
    values = F.linear(h, W["value_head.weight"])
    if training:
        return values[:, -1, :]
    eos = S - 1 - attention_mask.long().fliplr().argmax(dim=1)
    return values[torch.arange(B), eos, :]
