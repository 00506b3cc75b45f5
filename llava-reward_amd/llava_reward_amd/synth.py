"""Synthetic configs, deterministic weights and synthetic inputs for the reward-scoring path.

No checkpoint and no network exist in the build container or on the GPU box, so parity and
throughput are measured on seeded synthetic weights.  The generator is a
counter-based integer hash (splitmix64) so that three implementations produce bit-identical
tensors: this numpy one (tests, oracle, golden generation), the HIP kernel in
``csrc/rowops.hip`` (`synth_fill_kernel`; full-size weights generated straight into HBM) and any future port.

Element i of tensor `name` under base seed s:
    t   = splitmix64(s ^ fnv1a64(name))
    h   = splitmix64(t + i)
    v   = float(int(h >> 40) - 2**23) * scale          # one exactly-rounded fp32 multiply
    v   = offset + v ; optionally rounded to bf16 (RNE), as the reference loads bf16 checkpoints
          (eval/reward_adaptor_loader.py:33-40, torch_dtype=torch.bfloat16)
`scale = std * sqrt(12) / 2**24` gives a uniform distribution with the requested std.

Weight names follow the reference's state_dict (llava_reward/models/base_mllm/phi3_v/
modeling_phi3_v.py:1332-1374 Phi3VModel, :118-207 Phi3ImageEmbedding, transformers CLIPVisionModel;
llava_reward/models/rw_model_general_preference.py:314-326 for the reward heads).
"""
from __future__ import annotations

import dataclasses
import math
from typing import Dict, Iterator, List, Tuple

import numpy as np

MASK64 = (1 << 64) - 1


# ----------------------------------------------------------------------------------------------
# configuration
# ----------------------------------------------------------------------------------------------
@dataclasses.dataclass
class ClipConfig:
    """CLIP ViT geometry; the reference hard-wires ViT-L/14-336 (modeling_phi3_v.py:68-83) and
    uses hidden_states[-2], i.e. 23 of the 24 layers (utils/utils.py:264-282)."""
    hidden: int = 1024
    heads: int = 16
    mlp: int = 4096
    layers_used: int = 23
    image: int = 336
    patch: int = 14
    ln_eps: float = 1e-5

    @property
    def grid(self) -> int:
        return self.image // self.patch          # 24

    @property
    def tokens(self) -> int:
        return self.grid * self.grid + 1         # 577

    @property
    def head_dim(self) -> int:
        return self.hidden // self.heads


@dataclasses.dataclass
class RewardConfig:
    """Phi-3.5-V reward model geometry (configuration_phi3_v.py:119-145 + reward_config.yaml keys)."""
    vocab_size: int = 32064
    hidden: int = 3072
    intermediate: int = 8192
    layers: int = 32
    heads: int = 32
    rms_eps: float = 1e-5
    rope_theta: float = 10000.0
    max_pos: int = 131072
    orig_max_pos: int = 4096
    short_factor: Tuple[float, ...] = ()
    long_factor: Tuple[float, ...] = ()
    clip: ClipConfig = dataclasses.field(default_factory=ClipConfig)
    # reward head (eval/reward_adaptor_loader.py:25-30)
    is_general_preference: bool = False
    add_cross_attention: bool = True
    value_head_dim: int = 1
    general_preference_tau: float = 0.1
    ca_eps: float = 1e-5          # RMSNorm_class_eps passed by load_reward_adaptor (:32)
    # un-merged LoRA adapter on the decoder linears (eval/reward_adaptor_loader.py:44-45; scripts: --lora_rank 128 --lora_alpha 256)
    lora_rank: int = 0
    # su-RoPE switch point: False = eager / sdpa attention (long factors iff S > orig_max_pos, modeling_phi3_v.py:673), True =
    # Phi3FlashAttention2 (:793-794: iff S >= orig_max_pos) -- what the reference's --flash_attn scripts run
    rope_flash_convention: bool = False

    def __post_init__(self):
        if not self.short_factor:
            self.short_factor = default_rope_factors(self.head_dim, 1.0, 0.004)
        if not self.long_factor:
            self.long_factor = default_rope_factors(self.head_dim, 1.0, 0.8)
        if not self.is_general_preference:
            self.value_head_dim = 1

    @property
    def head_dim(self) -> int:
        return self.hidden // self.heads

    @property
    def proj_in(self) -> int:
        return 4 * self.clip.hidden

    def to_json(self) -> dict:
        d = dataclasses.asdict(self)
        d["short_factor"] = list(self.short_factor)
        d["long_factor"] = list(self.long_factor)
        return d

    @staticmethod
    def from_json(d: dict) -> "RewardConfig":
        d = dict(d)
        clip = ClipConfig(**d.pop("clip"))
        d["short_factor"] = tuple(d["short_factor"])
        d["long_factor"] = tuple(d["long_factor"])
        return RewardConfig(clip=clip, **d)


def default_rope_factors(head_dim: int, base: float, step: float) -> Tuple[float, ...]:
    """Synthetic su-RoPE factors (the real ones live in the checkpoint's config.json)."""
    return tuple(float(np.float32(base + step * i)) for i in range(head_dim // 2))


def full_config(**kw) -> RewardConfig:
    """Phi-3.5-vision-instruct shapes (BASELINE.json configs 1-3)."""
    return RewardConfig(**kw)


def ref_small_config(**kw) -> RewardConfig:
    """Full CLIP ViT-L (the reference cannot build any other tower) + a 2-layer, 384-wide LLM."""
    base = dict(vocab_size=1024, hidden=384, intermediate=512, layers=2, heads=4)
    base.update(kw)
    return RewardConfig(**base)


def tiny_config(**kw) -> RewardConfig:
    """Everything small (engine-vs-oracle tests; not buildable by the reference)."""
    base = dict(vocab_size=1024, hidden=384, intermediate=512, layers=2, heads=4,
                clip=ClipConfig(hidden=128, heads=2, mlp=512, layers_used=2))
    base.update(kw)
    return RewardConfig(**base)


# ----------------------------------------------------------------------------------------------
# counter-based RNG
# ----------------------------------------------------------------------------------------------
def fnv1a64(name: str) -> int:
    h = 0xCBF29CE484222325
    for b in name.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & MASK64
    return h


def splitmix64_scalar(x: int) -> int:
    x = (x + 0x9E3779B97F4A7C15) & MASK64
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
    return z ^ (z >> 31)


def tensor_seed(base_seed: int, name: str) -> int:
    return splitmix64_scalar((base_seed ^ fnv1a64(name)) & MASK64)


def _splitmix64_np(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def round_to_bf16_np(x: np.ndarray) -> np.ndarray:
    """fp32 -> nearest-even bf16, returned as fp32 (finite inputs only)."""
    u = x.view(np.uint32)
    r = (u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)
    return r.view(np.float32)


def uniform_scale(std: float) -> np.float32:
    return np.float32(std * math.sqrt(12.0) / 16777216.0)


def _gen_chunk(t: np.uint64, s: int, e: int, scale: np.float32, offset: float, out: np.ndarray) -> None:
    idx = np.arange(s, e, dtype=np.uint64)
    with np.errstate(over="ignore"):
        h = _splitmix64_np(t + idx)
    c = (h >> np.uint64(40)).astype(np.int64) - (1 << 23)
    v = c.astype(np.float32) * scale
    if offset != 0.0:
        v = v + np.float32(offset)
    out[s:e] = v


# Weight profiles (flags shared with lr_synth_weights_ex, include/llava_reward_hip.h).  PROFILE_FP32 = values not rounded to bf16
# (what a merged LoRA adapter leaves behind).  PROFILE_OUTLIER = the structure trained LLM checkpoints show and N(0, 0.02) init does
# not: three massive residual-stream channels (embedding columns and decoder down_proj rows x 200), norm gains of 2..30 on ~1.6 % of
# the channels, one element of 50 sigma and four elements below f16's normal range (x 2^-12) per matrix.  PROFILE_E4M3 = every
# matrix rounded (RNE) to the OCP e4m3 grid under one power-of-two scale per tensor: the weights of an fp8-weight checkpoint,
# de-quantised (BASELINE configs[4]); such values are exact in bf16 and f16.
PROFILE_FP32, PROFILE_OUTLIER, PROFILE_E4M3 = 1, 2, 4
PROFILE_NAMES = {"": 0, "default": 0, "fp32": PROFILE_FP32, "outlier": PROFILE_OUTLIER, "e4m3": PROFILE_E4M3}
_C_GAIN, _C_SPIKE, _C_TINY, _C_CHAN = 0xA5A5A5A55A5A5A5A, 0x0123456789ABCDEF, 0x0F1E2D3C4B5A6978, 0x5851F42D4C957F2D
OUTLIER_CHANNEL_SCALE, OUTLIER_SPIKE_SIGMAS, OUTLIER_N_CHANNELS, OUTLIER_N_TINY = 200.0, 50.0, 3, 4


def outlier_channels(base_seed: int, dim: int) -> List[int]:
    """The massive residual-stream channels of PROFILE_OUTLIER (one set per seed, shared by every tensor)."""
    return [int(splitmix64_scalar(((base_seed ^ _C_CHAN) + j) & MASK64) % dim) for j in range(OUTLIER_N_CHANNELS)]


def channel_axis(name: str) -> int:
    """0 = rows, 1 = columns of the tensor index the residual stream's channels, -1 = neither (PROFILE_OUTLIER)."""
    if name.endswith("embed_tokens.weight"):
        return 1
    if name.endswith(".mlp.down_proj.weight") and not name.startswith("visual."):
        return 0
    return -1


def _apply_outlier(v: np.ndarray, base_seed: int, name: str, rows: int, cols: int, std: float, offset: float) -> None:
    n = v.size
    t = tensor_seed(base_seed, name)
    if offset != 0.0:                                   # norm gain vectors
        with np.errstate(over="ignore"):
            h = _splitmix64_np(np.uint64(t ^ _C_GAIN) + np.arange(n, dtype=np.uint64))
        sel = (h & np.uint64(63)) == np.uint64(0)
        g = (np.uint64(2) + ((h >> np.uint64(8)) % np.uint64(29))).astype(np.float32)
        v[sel] = v[sel] * g[sel]
        return
    if rows <= 1 or cols <= 1 or std <= 0.0:
        return
    ax = channel_axis(name)
    if ax >= 0:
        m = v.reshape(rows, cols)
        for c in sorted(set(outlier_channels(base_seed, cols if ax == 1 else rows))):
            if ax == 1:
                m[:, c] = m[:, c] * np.float32(OUTLIER_CHANNEL_SCALE)
            else:
                m[c, :] = m[c, :] * np.float32(OUTLIER_CHANNEL_SCALE)
    for i in sorted({int(splitmix64_scalar(((t ^ _C_TINY) + k) & MASK64) % n) for k in range(OUTLIER_N_TINY)}):
        v[i] = v[i] * np.float32(2.0 ** -12)
    hs = splitmix64_scalar(t ^ _C_SPIKE)
    v[int(hs % n)] = np.float32(-1.0 if hs >> 63 else 1.0) * (np.float32(OUTLIER_SPIKE_SIGMAS) * np.float32(std))


def e4m3_tensor_exponent(std: float) -> int:
    """Smallest e with 448 * 2^e >= the largest magnitude gen_tensor can emit for `std` (2^23 * scale)."""
    bound = float(np.float32(8388608.0) * uniform_scale(std))
    m, ex = math.frexp(bound / 448.0)
    return ex - 1 if m == 0.5 else ex


def round_to_e4m3_np(v: np.ndarray, e: int) -> np.ndarray:
    """RNE onto the OCP e4m3 grid scaled by 2^e (normals: 4 significant bits; below 2^-6: steps of 2^-9; clamp at 448)."""
    x = np.clip(np.ldexp(v, -e), np.float32(-448.0), np.float32(448.0)).astype(np.float32)
    _, ex = np.frexp(x)
    qe = np.maximum(ex - 4, -9).astype(np.int32)
    q = np.ldexp(np.rint(np.ldexp(x, -qe)), qe).astype(np.float32)
    return np.ldexp(q, e).astype(np.float32)


def gen_tensor(base_seed: int, name: str, shape: Tuple[int, ...], std: float, offset: float = 0.0,
               bf16_valued: bool = True, chunk: int = 1 << 22, profile: int = 0) -> np.ndarray:
    """Deterministic fp32 tensor; identical to csrc/rowops.hip synth_fill_kernel (+ synth_profile_kernel) for the same arguments."""
    n = int(np.prod(shape)) if len(shape) else 1
    out = np.empty(n, dtype=np.float32)
    t = np.uint64(tensor_seed(base_seed, name))
    scale = uniform_scale(std)
    spans = [(s, min(n, s + chunk)) for s in range(0, n, chunk)]
    if len(spans) > 1:
        from concurrent.futures import ThreadPoolExecutor
        import os
        with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
            list(ex.map(lambda se: _gen_chunk(t, se[0], se[1], scale, offset, out), spans))
    else:
        for s, e in spans:
            _gen_chunk(t, s, e, scale, offset, out)
    rows = int(shape[0]) if len(shape) > 1 else 1
    cols = n // max(rows, 1)
    if profile & PROFILE_OUTLIER:
        _apply_outlier(out, base_seed, name, rows, cols, std, offset)
    if profile & PROFILE_E4M3 and rows > 1 and cols > 1 and std > 0.0 and offset == 0.0:
        out = round_to_e4m3_np(out, e4m3_tensor_exponent(std))
    if bf16_valued and not (profile & PROFILE_FP32):
        out = round_to_bf16_np(out)
    return out.reshape(shape)


# ----------------------------------------------------------------------------------------------
# weight inventory
# ----------------------------------------------------------------------------------------------
CLIP_PREFIX = "model.vision_embed_tokens.img_processor.vision_model."
EMB_PREFIX = "model.vision_embed_tokens."


def lora_specs(mod: str, n_out: int, n_in: int, rank: int) -> List[Tuple[str, Tuple[int, ...], float, float]]:
    """The two tensors of an un-merged adapter on linear `mod` (peft LoraLayer: lora_A [r, in], lora_B [out, r]).  The engine
    takes lora_B PRE-SCALED by lora_alpha / r (include/llava_reward_hip.h `lora_rank`); synthetic tensors are that product."""
    return [(mod + ".lora_A.weight", (rank, n_in), 0.02, 0.0), (mod + ".lora_B.weight", (n_out, rank), 0.02, 0.0)]


def weight_specs(cfg: RewardConfig) -> List[Tuple[str, Tuple[int, ...], float, float]]:
    """(name, shape, std, offset) for every tensor the path reads.  Linear/embedding std 0.02
    mirrors _init_weights (modeling_phi3_v.py:1241-1250); norm weights are 1 + small noise so
    that a dropped or mis-ordered scale vector is visible to the parity tests."""
    c = cfg.clip
    D, I = cfg.hidden, cfg.intermediate
    s: List[Tuple[str, Tuple[int, ...], float, float]] = []
    s.append(("model.embed_tokens.weight", (cfg.vocab_size, D), 0.02, 0.0))
    # CLIP tower
    s.append((CLIP_PREFIX + "embeddings.class_embedding", (c.hidden,), 0.02, 0.0))
    s.append((CLIP_PREFIX + "embeddings.patch_embedding.weight", (c.hidden, 3, c.patch, c.patch), 0.02, 0.0))
    s.append((CLIP_PREFIX + "embeddings.position_embedding.weight", (c.tokens, c.hidden), 0.02, 0.0))
    s.append((CLIP_PREFIX + "pre_layrnorm.weight", (c.hidden,), 0.05, 1.0))
    s.append((CLIP_PREFIX + "pre_layrnorm.bias", (c.hidden,), 0.02, 0.0))
    for l in range(c.layers_used):
        p = f"{CLIP_PREFIX}encoder.layers.{l}."
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            s.append((p + f"self_attn.{nm}.weight", (c.hidden, c.hidden), 0.02, 0.0))
            s.append((p + f"self_attn.{nm}.bias", (c.hidden,), 0.02, 0.0))
        s.append((p + "layer_norm1.weight", (c.hidden,), 0.05, 1.0))
        s.append((p + "layer_norm1.bias", (c.hidden,), 0.02, 0.0))
        s.append((p + "mlp.fc1.weight", (c.mlp, c.hidden), 0.02, 0.0))
        s.append((p + "mlp.fc1.bias", (c.mlp,), 0.02, 0.0))
        s.append((p + "mlp.fc2.weight", (c.hidden, c.mlp), 0.02, 0.0))
        s.append((p + "mlp.fc2.bias", (c.hidden,), 0.02, 0.0))
        s.append((p + "layer_norm2.weight", (c.hidden,), 0.05, 1.0))
        s.append((p + "layer_norm2.bias", (c.hidden,), 0.02, 0.0))
    # HD transform separators + projector (modeling_phi3_v.py:165-179)
    s.append((EMB_PREFIX + "glb_GN", (1, 1, cfg.proj_in), 0.02, 0.0))
    s.append((EMB_PREFIX + "sub_GN", (1, 1, 1, cfg.proj_in), 0.02, 0.0))
    s.append((EMB_PREFIX + "img_projection.0.weight", (D, cfg.proj_in), 0.02, 0.0))
    s.append((EMB_PREFIX + "img_projection.0.bias", (D,), 0.02, 0.0))
    s.append((EMB_PREFIX + "img_projection.2.weight", (D, D), 0.02, 0.0))
    s.append((EMB_PREFIX + "img_projection.2.bias", (D,), 0.02, 0.0))
    # decoder
    for l in range(cfg.layers):
        p = f"model.layers.{l}."
        s.append((p + "input_layernorm.weight", (D,), 0.05, 1.0))
        s.append((p + "self_attn.qkv_proj.weight", (3 * D, D), 0.02, 0.0))
        s.append((p + "self_attn.o_proj.weight", (D, D), 0.02, 0.0))
        s.append((p + "post_attention_layernorm.weight", (D,), 0.05, 1.0))
        s.append((p + "mlp.gate_up_proj.weight", (2 * I, D), 0.02, 0.0))
        s.append((p + "mlp.down_proj.weight", (D, I), 0.02, 0.0))
        if cfg.lora_rank > 0:           # utils/utils.py:194-222 (LLM targets; the recipes freeze the vision tower)
            s += lora_specs(p + "self_attn.qkv_proj", 3 * D, D, cfg.lora_rank)
            s += lora_specs(p + "self_attn.o_proj", D, D, cfg.lora_rank)
            s += lora_specs(p + "mlp.gate_up_proj", 2 * I, D, cfg.lora_rank)
            s += lora_specs(p + "mlp.down_proj", D, I, cfg.lora_rank)
    s.append(("model.norm.weight", (D,), 0.05, 1.0))
    # reward heads (rw_model_general_preference.py:314-326)
    if cfg.add_cross_attention:
        for nm in ("W_q", "W_k", "W_v"):
            s.append((nm + ".weight", (D, D), 0.02, 0.0))
        s.append(("ca_layernorm.weight", (D,), 0.05, 1.0))
    s.append(("value_head.weight", (cfg.value_head_dim, D), 1.0 / math.sqrt(D), 0.0))
    return s


def iter_weights(cfg: RewardConfig, seed: int, profile: int = 0) -> Iterator[Tuple[str, np.ndarray]]:
    for name, shape, std, offset in weight_specs(cfg):
        yield name, gen_tensor(seed, name, shape, std, offset, profile=profile)


def make_weights(cfg: RewardConfig, seed: int, profile: int = 0) -> Dict[str, np.ndarray]:
    return dict(iter_weights(cfg, seed, profile))


# ----------------------------------------------------------------------------------------------
# input geometry (restates processing_phi3_v.py:83-136,194,269-272 token/crop arithmetic)
# ----------------------------------------------------------------------------------------------
def hd_target_size(width: int, height: int, num_crops: int = 16) -> Tuple[int, int, bool]:
    """Size the HD transform resizes+pads to, as (h, w, transposed).  processing_phi3_v.py:83-104."""
    trans = False
    if width < height:
        width, height = height, width
        trans = True
    ratio = width / height
    scale = 1
    while scale * math.ceil(scale / ratio) <= num_crops:
        scale += 1
    scale -= 1
    new_w = int(scale * 336)
    new_h = int(new_w / ratio)
    # padding_336 (processing_phi3_v.py:62-71): height padded up to a multiple of 336
    tar = int(math.ceil(new_h / 336) * 336)
    if trans:
        return new_w, tar, True       # image is transposed back: (h, w) = (new_w, padded h)
    return tar, new_w, False


def num_img_tokens(h: int, w: int) -> int:
    """processing_phi3_v.py:269: ((h/336)*(w/336)+1)*144 + 1 + (h/336+1)*12."""
    return int((h // 336) * (w // 336) + 1) * 144 + 1 + int(h // 336 + 1) * 12


def synth_pixels(seed: int, name: str, shape: Tuple[int, ...]) -> np.ndarray:
    """CLIP-normalised-like pixel noise, std 1, fp32 (not bf16-valued)."""
    return gen_tensor(seed, name, shape, 1.0, 0.0, bf16_valued=False)


def synth_image(seed: int, name: str, h: int, w: int, smooth: bool = False) -> np.ndarray:
    """Seeded RGB uint8 [h, w, 3] test image: byte noise, or (smooth) gradients + noise in the low 3 bits, the kind of
    content whose resampled bytes land on rounding boundaries."""
    t = np.uint64(tensor_seed(seed, name))
    idx = np.arange(h * w * 3, dtype=np.uint64)
    noise = (_splitmix64_np(t + idx) >> np.uint64(56)).astype(np.uint8).reshape(h, w, 3)
    if not smooth:
        return noise
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([yy * 255 // max(h - 1, 1), xx * 255 // max(w - 1, 1), (yy + 2 * xx) % 256], axis=-1)
    return ((base & 0xF8) | (noise & 7)).astype(np.uint8)


def pad_left(batch: Dict[str, np.ndarray], k: int, pad_token_id: int = None) -> Dict[str, np.ndarray]:
    """k more LEFT-padding columns on every row of a synth_batch (mask 0): a collated batch never looks like this (its longest
    row is un-padded) but custom_forward accepts it, and it is the case where `S` and `max(position_ids) + 1` differ."""
    if k <= 0:
        return batch
    ids, mask = batch["input_ids"], batch["attention_mask"]
    pad = ids[mask == 0][0] if (mask == 0).any() and pad_token_id is None else (0 if pad_token_id is None else pad_token_id)
    out = dict(batch)
    out["input_ids"] = np.concatenate([np.full((ids.shape[0], k), pad, dtype=ids.dtype), ids], axis=1)
    out["attention_mask"] = np.concatenate([np.zeros((ids.shape[0], k), dtype=mask.dtype), mask], axis=1)
    return out


def right_pad(batch: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """The same rows with their padding moved to the RIGHT (valid tokens first): what a right-padding tokenizer would collate."""
    ids, mask = batch["input_ids"], batch["attention_mask"]
    out = dict(batch)
    out["input_ids"], out["attention_mask"] = ids.copy(), mask.copy()
    for b in range(ids.shape[0]):
        v = mask[b] == 1
        n = int(v.sum())
        out["input_ids"][b] = np.concatenate([ids[b][v], ids[b][~v]])
        out["attention_mask"][b] = np.concatenate([np.ones(n, dtype=mask.dtype), np.zeros(len(v) - n, dtype=mask.dtype)])
    return out


class StandInTokenizer:
    """Deterministic stand-in for the Phi-3.5-V tokenizer (no tokenizer files exist offline) with the three members
    inference_process_phi3v uses (eval/reward_adaptor_loader.py:163-167): `apply_chat_template`, `eos_token`, `__call__`.
    One token per character, ids in [3, 203); the 22 characters the caller strips are the real template's
    `<|end|>\n<|assistant|>\n`.  Used by tests/golden/make_goldens.py `pair_sample` and the GPU test that replays it."""
    eos_token = "<|endoftext|>"
    pad_token = "<|endoftext|>"
    pad_token_id = 32000
    eos_token_id = 32000
    padding_side = "left"

    def apply_chat_template(self, messages, tokenize=False, add_generation_prompt=True):
        return "<|user|>\n" + messages[0]["content"] + "<|end|>\n<|assistant|>\n"

    def __call__(self, text):
        class _Enc:
            pass
        r = _Enc()
        r.input_ids = [3 + (ord(ch) % 200) for ch in text]
        return r


# the caption of the reference's own sample pair (eval/simple_inference.py:21, data/sample_test/pairwise_sample.json): input data
SAMPLE_CAPTION = ("perfect white haired egyptian goddess wearing white dove wings, warframe armor, regal, attractive, ornate, sultry, "
                  "beautiful, ice queen, half asian, pretty face, blue eyes, detailed, scifi platform, 4 k, ultra realistic, epic "
                  "lighting, illuminated, cinematic, masterpiece, art by akihito tsukushi, voidstar")


def synth_batch(cfg: RewardConfig, seed: int, caption_lens: List[int], grids, max_crops: int = None,
                pad_token_id: int = None, with_pixels: bool = True):
    """Token/mask/pixel tensors shaped like collate_fn output (reward_dataset.py:137-202, Appendix B
    of SURVEY.md): each row is [bos, <|user|>, \\n, -1 x V_b, \\n, caption..., eos]; rows are LEFT-padded
    with pad_token_id / mask 0 to the longest row (datasets/utils.py:5-13).
    `grids` is one (h_crop, w_crop) for all samples or a list with one per sample.
    Returns numpy arrays: input_ids [B,S] i64, attention_mask [B,S] i64,
    pixel_values [B,C,3,336,336] f32 (crop 0 = global, then local crops row-major, zero-padded to C,
    processing_phi3_v.py:272-278), image_sizes [B,2] i64 (HD-transformed h, w)."""
    batch = len(caption_lens)
    if isinstance(grids[0], int):
        grids = [tuple(grids)] * batch
    assert len(grids) == batch
    ncrops = [hc * wc + 1 for hc, wc in grids]
    C = max_crops + 1 if max_crops is not None else max(ncrops)
    assert C >= max(ncrops)
    hi = min(32000, cfg.vocab_size - 1)
    pad_id = pad_token_id if pad_token_id is not None else hi
    lo = 3
    rows = []
    for b, n in enumerate(caption_lens):
        V = num_img_tokens(336 * grids[b][0], 336 * grids[b][1])
        t = tensor_seed(seed, f"caption.{b}")
        cap = np.array([lo + splitmix64_scalar((t + i) & MASK64) % (hi - lo) for i in range(n)],
                       dtype=np.int64)
        rows.append(np.concatenate([np.array([1, 2, 13], dtype=np.int64), np.full(V, -1, dtype=np.int64),
                                    np.array([13], dtype=np.int64), cap, np.array([pad_id], dtype=np.int64)]))
    S = max(len(r) for r in rows)
    ids = np.full((batch, S), pad_id, dtype=np.int64)
    mask = np.zeros((batch, S), dtype=np.int64)
    for b, row in enumerate(rows):
        ids[b, S - len(row):] = row
        mask[b, S - len(row):] = 1
    pix = None
    if with_pixels:
        pix = np.zeros((batch, C, 3, 336, 336), dtype=np.float32)
        for b in range(batch):
            pix[b, :ncrops[b]] = synth_pixels(seed, f"pixel_values.{b}", (ncrops[b], 3, 336, 336))
    sizes = np.array([[336 * hc, 336 * wc] for hc, wc in grids], dtype=np.int64)
    return dict(input_ids=ids, attention_mask=mask, pixel_values=pix, image_sizes=sizes)


# ----------------------------------------------------------------------------------------------
# LLaVA-1.6 (LlavaNext + Mistral) reward model: rw_model_general_preference.py:372-375 branch
# ----------------------------------------------------------------------------------------------
LLAVA_PINPOINTS = ((336, 672), (672, 336), (672, 672), (1008, 336), (336, 1008))


@dataclasses.dataclass
class LlavaConfig:
    """llava-hf/llava-v1.6-mistral-7b-hf geometry (public config.json; not in the reference tree)."""
    vocab_size: int = 32064
    hidden: int = 4096
    intermediate: int = 14336
    layers: int = 32
    heads: int = 32
    kv_heads: int = 8
    head_dim: int = 128
    rms_eps: float = 1e-5
    rope_theta: float = 1000000.0
    clip: ClipConfig = dataclasses.field(default_factory=ClipConfig)
    image_token_id: int = 32000
    pad_token_id: int = 32001
    pinpoints: Tuple[Tuple[int, int], ...] = LLAVA_PINPOINTS
    is_general_preference: bool = False
    add_cross_attention: bool = False          # the llava branch has no SkipCA (rw_model:376-397)
    value_head_dim: int = 1
    general_preference_tau: float = 0.1
    lora_rank: int = 0                         # un-merged adapter on q/k/v/o/gate/up/down of every layer (utils/utils.py:243-262)

    def __post_init__(self):
        if not self.is_general_preference:
            self.value_head_dim = 1
        if self.add_cross_attention:
            raise ValueError("add_cross_attention=True raises in the reference for llava (rw_model:315: no config.hidden_size)")

    def to_json(self) -> dict:
        d = dataclasses.asdict(self)
        d["pinpoints"] = [list(p) for p in self.pinpoints]
        d["backbone"] = "llava"
        return d

    @staticmethod
    def from_json(d: dict) -> "LlavaConfig":
        d = dict(d)
        d.pop("backbone", None)
        clip = ClipConfig(**d.pop("clip"))
        d["pinpoints"] = tuple(tuple(p) for p in d["pinpoints"])
        return LlavaConfig(clip=clip, **d)


def llava_full_config(**kw) -> LlavaConfig:
    return LlavaConfig(**kw)


def llava_tiny_config(**kw) -> LlavaConfig:
    base = dict(vocab_size=1024, hidden=512, intermediate=768, layers=2, heads=4, kv_heads=2, head_dim=128,
                image_token_id=1000, pad_token_id=1001, clip=ClipConfig(hidden=128, heads=2, mlp=512, layers_used=2))
    base.update(kw)
    return LlavaConfig(**base)


def select_best_resolution(original_size, possible_resolutions):
    """transformers.image_processing_utils.select_best_resolution (third party, restated)."""
    oh, ow = original_size
    best, max_eff, min_waste = None, 0, float("inf")
    for h, w in possible_resolutions:
        scale = min(w / ow, h / oh)
        dw, dh = int(ow * scale), int(oh * scale)
        eff = min(dw * dh, ow * oh)
        waste = w * h - eff
        if eff > max_eff or (eff == max_eff and waste < min_waste):
            max_eff, min_waste, best = eff, waste, (h, w)
    return best


def llava_geometry(h: int, w: int, pinpoints=LLAVA_PINPOINTS, crop: int = 336, g: int = 24):
    """(grid_h, grid_w, r0, r1, c0, c1, n_tokens) of one image of original size (h, w): the anyres grid picked by
    select_best_resolution, the rows/cols of the (grid_h*g) x (grid_w*g) feature map that survive unpad_image
    (modeling_llava_next.py:109-146) and the packed token count = g*g + rows*(cols+1)."""
    bh, bw = select_best_resolution((h, w), pinpoints)
    gh, gw = bh // crop, bw // crop
    ch, cw = gh * g, gw * g
    r0, r1, c0, c1 = 0, ch, 0, cw
    if w / h > cw / ch:
        nh = int(round(h * (cw / w), 7))
        pad = (ch - nh) // 2
        r0, r1 = pad, ch - pad
    else:
        nw = int(round(w * (ch / h), 7))
        pad = (cw - nw) // 2
        c0, c1 = pad, cw - pad
    return gh, gw, r0, r1, c0, c1, g * g + (r1 - r0) * (c1 - c0 + 1)


LLAVA_CLIP_PREFIX = "vision_tower.vision_model."


def llava_weight_specs(cfg: LlavaConfig) -> List[Tuple[str, Tuple[int, ...], float, float]]:
    """Checkpoint (transformers 4.50 era) names of llava-v1.6-mistral-7b-hf + the reward head."""
    c = cfg.clip
    D, I, hd = cfg.hidden, cfg.intermediate, cfg.head_dim
    s: List[Tuple[str, Tuple[int, ...], float, float]] = []
    s.append(("language_model.model.embed_tokens.weight", (cfg.vocab_size, D), 0.02, 0.0))
    p = LLAVA_CLIP_PREFIX
    s.append((p + "embeddings.class_embedding", (c.hidden,), 0.02, 0.0))
    s.append((p + "embeddings.patch_embedding.weight", (c.hidden, 3, c.patch, c.patch), 0.02, 0.0))
    s.append((p + "embeddings.position_embedding.weight", (c.tokens, c.hidden), 0.02, 0.0))
    s.append((p + "pre_layrnorm.weight", (c.hidden,), 0.05, 1.0))
    s.append((p + "pre_layrnorm.bias", (c.hidden,), 0.02, 0.0))
    for l in range(c.layers_used):
        q = f"{p}encoder.layers.{l}."
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            s.append((q + f"self_attn.{nm}.weight", (c.hidden, c.hidden), 0.02, 0.0))
            s.append((q + f"self_attn.{nm}.bias", (c.hidden,), 0.02, 0.0))
        s.append((q + "layer_norm1.weight", (c.hidden,), 0.05, 1.0))
        s.append((q + "layer_norm1.bias", (c.hidden,), 0.02, 0.0))
        s.append((q + "mlp.fc1.weight", (c.mlp, c.hidden), 0.02, 0.0))
        s.append((q + "mlp.fc1.bias", (c.mlp,), 0.02, 0.0))
        s.append((q + "mlp.fc2.weight", (c.hidden, c.mlp), 0.02, 0.0))
        s.append((q + "mlp.fc2.bias", (c.hidden,), 0.02, 0.0))
        s.append((q + "layer_norm2.weight", (c.hidden,), 0.05, 1.0))
        s.append((q + "layer_norm2.bias", (c.hidden,), 0.02, 0.0))
    s.append(("multi_modal_projector.linear_1.weight", (D, c.hidden), 0.02, 0.0))
    s.append(("multi_modal_projector.linear_1.bias", (D,), 0.02, 0.0))
    s.append(("multi_modal_projector.linear_2.weight", (D, D), 0.02, 0.0))
    s.append(("multi_modal_projector.linear_2.bias", (D,), 0.02, 0.0))
    s.append(("image_newline", (D,), 0.02, 0.0))
    for l in range(cfg.layers):
        q = f"language_model.model.layers.{l}."
        s.append((q + "input_layernorm.weight", (D,), 0.05, 1.0))
        s.append((q + "self_attn.q_proj.weight", (cfg.heads * hd, D), 0.02, 0.0))
        s.append((q + "self_attn.k_proj.weight", (cfg.kv_heads * hd, D), 0.02, 0.0))
        s.append((q + "self_attn.v_proj.weight", (cfg.kv_heads * hd, D), 0.02, 0.0))
        s.append((q + "self_attn.o_proj.weight", (D, cfg.heads * hd), 0.02, 0.0))
        s.append((q + "post_attention_layernorm.weight", (D,), 0.05, 1.0))
        s.append((q + "mlp.gate_proj.weight", (I, D), 0.02, 0.0))
        s.append((q + "mlp.up_proj.weight", (I, D), 0.02, 0.0))
        s.append((q + "mlp.down_proj.weight", (D, I), 0.02, 0.0))
        if cfg.lora_rank > 0:
            for nm, n_out, n_in in (("self_attn.q_proj", cfg.heads * hd, D), ("self_attn.k_proj", cfg.kv_heads * hd, D),
                                    ("self_attn.v_proj", cfg.kv_heads * hd, D), ("self_attn.o_proj", D, cfg.heads * hd),
                                    ("mlp.gate_proj", I, D), ("mlp.up_proj", I, D), ("mlp.down_proj", D, I)):
                s += lora_specs(q + nm, n_out, n_in, cfg.lora_rank)
    s.append(("language_model.model.norm.weight", (D,), 0.05, 1.0))
    s.append(("value_head.weight", (cfg.value_head_dim, D), 1.0 / math.sqrt(D), 0.0))
    return s


def llava_make_weights(cfg: LlavaConfig, seed: int, profile: int = 0) -> Dict[str, np.ndarray]:
    return {n: gen_tensor(seed, n, sh, std, off, profile=profile) for n, sh, std, off in llava_weight_specs(cfg)}


def llava_synth_batch(cfg: LlavaConfig, seed: int, caption_lens: List[int], image_sizes, max_crops: int = None,
                      with_pixels: bool = True):
    """inputs_batch of the llava branch: input_ids (image slots = image_token_id, already expanded by the
    processor), attention_mask (left padded), pixel_values [B, P, 3, 336, 336] (crop 0 = base, zero-padded to P),
    image_sizes [B, 2] = ORIGINAL (h, w)."""
    batch = len(caption_lens)
    geo = [llava_geometry(int(h), int(w), cfg.pinpoints) for h, w in image_sizes]
    ncrops = [1 + g[0] * g[1] for g in geo]
    P = max_crops if max_crops is not None else max(ncrops)
    lo, hi = 3, min(cfg.image_token_id, cfg.pad_token_id) - 1
    rows = []
    for b, n in enumerate(caption_lens):
        t = tensor_seed(seed, f"caption.{b}")
        cap = np.array([lo + splitmix64_scalar((t + i) & MASK64) % (hi - lo) for i in range(n)], dtype=np.int64)
        rows.append(np.concatenate([np.array([1, 5, 6], dtype=np.int64), np.full(geo[b][6], cfg.image_token_id, dtype=np.int64),
                                    np.array([13], dtype=np.int64), cap, np.array([2], dtype=np.int64)]))
    S = max(len(r) for r in rows)
    ids = np.full((batch, S), cfg.pad_token_id, dtype=np.int64)
    mask = np.zeros((batch, S), dtype=np.int64)
    for b, row in enumerate(rows):
        ids[b, S - len(row):] = row
        mask[b, S - len(row):] = 1
    pix = None
    if with_pixels:
        pix = np.zeros((batch, P, 3, 336, 336), dtype=np.float32)
        for b in range(batch):
            pix[b, :ncrops[b]] = synth_pixels(seed, f"pixel_values.{b}", (ncrops[b], 3, 336, 336))
    sizes = np.array([[int(h), int(w)] for h, w in image_sizes], dtype=np.int64)
    return dict(input_ids=ids, attention_mask=mask, pixel_values=pix, image_sizes=sizes)


# ----------------------------------------------------------------------------------------------
# Qwen2.5-VL reward model: rw_model_general_preference.py:354-371 + :387-397 branch
# ----------------------------------------------------------------------------------------------
QWEN_CA_TOKEN_ID = 151643      # hard-wired at rw_model_general_preference.py:358 (<|endoftext|>, the pad token)


@dataclasses.dataclass
class QwenVisionConfig:
    """Qwen2.5-VL ViT geometry (public Qwen2.5-VL-7B-Instruct config.json `vision_config`; not in the reference tree)."""
    depth: int = 32
    hidden: int = 1280
    heads: int = 16
    intermediate: int = 3420
    patch: int = 14
    temporal_patch: int = 2
    merge: int = 2
    window: int = 112
    fullatt: Tuple[int, ...] = (7, 15, 23, 31)
    in_ch: int = 3
    rope_theta: float = 10000.0
    eps: float = 1e-6

    @property
    def head_dim(self) -> int:
        return self.hidden // self.heads

    @property
    def patch_dim(self) -> int:
        return self.in_ch * self.temporal_patch * self.patch * self.patch

    @property
    def merge_unit(self) -> int:
        return self.merge * self.merge

    @property
    def window_merged(self) -> int:
        """Side of an attention window in merged (LLM) tokens: window // merge // patch."""
        return self.window // self.merge // self.patch


@dataclasses.dataclass
class QwenConfig:
    """Qwen2.5-VL-7B-Instruct geometry (public config.json; BASELINE.json config #4) + the reward head keys."""
    vocab_size: int = 152064
    hidden: int = 3584
    intermediate: int = 18944
    layers: int = 28
    heads: int = 28
    kv_heads: int = 4
    head_dim: int = 128
    rms_eps: float = 1e-6
    rope_theta: float = 1000000.0
    mrope_section: Tuple[int, int, int] = (16, 24, 24)
    vision: QwenVisionConfig = dataclasses.field(default_factory=QwenVisionConfig)
    image_token_id: int = 151655      # <|image_pad|>
    pad_token_id: int = 151643        # <|endoftext|>; left padding (utils/utils.py:40-42)
    is_general_preference: bool = False
    add_cross_attention: bool = True
    value_head_dim: int = 1
    general_preference_tau: float = 0.1
    ca_eps: float = 1e-6              # RMSNorm_class_eps for qwen (eval/reward_adaptor_loader.py:68)
    lora_rank: int = 0                # un-merged adapter on q/k/v/o/gate/up/down of every decoder layer (utils/utils.py:223-242)

    def __post_init__(self):
        if not self.is_general_preference:
            self.value_head_dim = 1
        assert sum(self.mrope_section) * 2 == self.head_dim
        assert self.heads * self.head_dim == self.hidden, "Qwen2_5_VLAttention requires heads*head_dim == hidden"

    def to_json(self) -> dict:
        d = dataclasses.asdict(self)
        d["mrope_section"] = list(self.mrope_section)
        d["vision"]["fullatt"] = list(self.vision.fullatt)
        d["backbone"] = "qwen"
        return d

    @staticmethod
    def from_json(d: dict) -> "QwenConfig":
        d = dict(d)
        d.pop("backbone", None)
        v = dict(d.pop("vision"))
        v["fullatt"] = tuple(v["fullatt"])
        d["mrope_section"] = tuple(d["mrope_section"])
        return QwenConfig(vision=QwenVisionConfig(**v), **d)


def qwen_full_config(**kw) -> QwenConfig:
    return QwenConfig(**kw)


def qwen_tiny_config(**kw) -> QwenConfig:
    """Small everything; vocab 1024 means no token can equal 151643, so SkipCA sees no rows (attn_o = 0)."""
    base = dict(vocab_size=1024, hidden=512, intermediate=768, layers=2, heads=4, kv_heads=2, head_dim=128,
                image_token_id=1000, pad_token_id=1001,
                vision=QwenVisionConfig(depth=3, hidden=320, heads=4, intermediate=200, fullatt=(1,)))
    base.update(kw)
    return QwenConfig(**base)


def qwen_quirk_config(**kw) -> QwenConfig:
    """Tiny model with the real vocabulary ids, so that left padding (151643) feeds the as-written SkipCA."""
    base = dict(vocab_size=151680, hidden=256, heads=2, kv_heads=1, image_token_id=151655, pad_token_id=151643)
    base.update(kw)
    return qwen_tiny_config(**base)


def qwen_window_index(grid_thw, vc: QwenVisionConfig):
    """transformers.vision_utils.get_vision_window_index (third party, restated): the permutation that groups
    merge units (2x2 patches) into window-major order, and the cumulative TOKEN counts of the windows."""
    ws, unit = vc.window_merged, vc.merge_unit
    index, cu, base = [], [0], 0
    for t, h, w in grid_thw:
        gh, gw = h // vc.merge, w // vc.merge
        idx = np.arange(t * gh * gw, dtype=np.int64).reshape(t, gh, gw)
        ph, pw = ws - gh % ws, ws - gw % ws            # a full extra window when divisible: it stays empty
        nh, nw = (gh + ph) // ws, (gw + pw) // ws
        pad = np.full((t, gh + ph, gw + pw), -100, dtype=np.int64)
        pad[:, :gh, :gw] = idx
        pad = pad.reshape(t, nh, ws, nw, ws).transpose(0, 1, 3, 2, 4).reshape(t, nh * nw, ws, ws)
        lens = (pad != -100).sum(axis=(2, 3)).reshape(-1)
        flat = pad.reshape(-1)
        index.append(flat[flat != -100] + base)
        for n in lens:
            cu.append(cu[-1] + int(n) * unit)
        base += t * gh * gw
    cu_u = [cu[0]]
    for c in cu[1:]:
        if c != cu_u[-1]:
            cu_u.append(c)
    return np.concatenate(index), np.array(cu_u, dtype=np.int64)


def qwen_patch_positions(grid_thw, vc: QwenVisionConfig) -> np.ndarray:
    """transformers.vision_utils.get_vision_position_ids: (h, w) of every patch, in the processor's patch order
    (merge-block major: [h/m, w/m, m, m])."""
    out = []
    m = vc.merge
    for t, h, w in grid_thw:
        hp, wp = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
        hp = hp.reshape(h // m, m, w // m, m).transpose(0, 2, 1, 3).reshape(-1)
        wp = wp.reshape(h // m, m, w // m, m).transpose(0, 2, 1, 3).reshape(-1)
        out.append(np.tile(np.stack([hp, wp], axis=-1), (t, 1)))
    return np.concatenate(out).astype(np.int64)


def qwen_rope_index(input_ids: np.ndarray, attention_mask: np.ndarray, grid_thw, cfg: QwenConfig) -> np.ndarray:
    """Qwen2_5_VLModel.get_rope_index for still images (third party, restated): [3, B, S] (t, h, w) positions.
    Text runs count up from the running position; an image run gets t = start, h/w = start + merged-grid
    coordinates, and advances the running position by max(h, w) / merge.  Masked positions stay 0."""
    B, S = input_ids.shape
    pos = np.zeros((3, B, S), dtype=np.int64)
    it = iter(grid_thw)
    m = cfg.vision.merge
    for b in range(B):
        keep = np.nonzero(attention_mask[b])[0]
        ids = input_ids[b, keep]
        out = np.zeros((3, len(ids)), dtype=np.int64)
        cur, i = 0, 0
        while i < len(ids):
            is_img = ids[i] == cfg.image_token_id
            j = i
            while j < len(ids) and (ids[j] == cfg.image_token_id) == is_img:
                j += 1
            if not is_img:
                out[:, i:j] = cur + np.arange(j - i)
                cur += j - i
            else:
                t, h, w = next(it)
                gh, gw = h // m, w // m
                assert j - i == t * gh * gw, "image slot run does not match image_grid_thw"
                tt, hh, ww = np.meshgrid(np.arange(t), np.arange(gh), np.arange(gw), indexing="ij")
                out[0, i:j] = tt.reshape(-1) + cur
                out[1, i:j] = hh.reshape(-1) + cur
                out[2, i:j] = ww.reshape(-1) + cur
                cur += max(h, w) // m
            i = j
        pos[:, b, keep] = out
    return pos


def qwen_weight_specs(cfg: QwenConfig) -> List[Tuple[str, Tuple[int, ...], float, float]]:
    """Checkpoint (transformers 4.50 era) names of Qwen2.5-VL-*-Instruct + the reward head.  lm_head is
    omitted: the reference computes the logits and never reads them (rw_model:357)."""
    v = cfg.vision
    D, I, hd = cfg.hidden, cfg.intermediate, cfg.head_dim
    s: List[Tuple[str, Tuple[int, ...], float, float]] = []
    s.append(("model.embed_tokens.weight", (cfg.vocab_size, D), 0.02, 0.0))
    s.append(("visual.patch_embed.proj.weight", (v.hidden, v.in_ch, v.temporal_patch, v.patch, v.patch), 0.02, 0.0))
    for l in range(v.depth):
        q = f"visual.blocks.{l}."
        s.append((q + "norm1.weight", (v.hidden,), 0.05, 1.0))
        s.append((q + "attn.qkv.weight", (3 * v.hidden, v.hidden), 0.02, 0.0))
        s.append((q + "attn.qkv.bias", (3 * v.hidden,), 0.02, 0.0))
        s.append((q + "attn.proj.weight", (v.hidden, v.hidden), 0.02, 0.0))
        s.append((q + "attn.proj.bias", (v.hidden,), 0.02, 0.0))
        s.append((q + "norm2.weight", (v.hidden,), 0.05, 1.0))
        for nm, sh in (("gate_proj", (v.intermediate, v.hidden)), ("up_proj", (v.intermediate, v.hidden)),
                       ("down_proj", (v.hidden, v.intermediate))):
            s.append((q + f"mlp.{nm}.weight", sh, 0.02, 0.0))
            s.append((q + f"mlp.{nm}.bias", (sh[0],), 0.02, 0.0))
    mh = v.hidden * v.merge_unit
    s.append(("visual.merger.ln_q.weight", (v.hidden,), 0.05, 1.0))
    s.append(("visual.merger.mlp.0.weight", (mh, mh), 0.02, 0.0))
    s.append(("visual.merger.mlp.0.bias", (mh,), 0.02, 0.0))
    s.append(("visual.merger.mlp.2.weight", (D, mh), 0.02, 0.0))
    s.append(("visual.merger.mlp.2.bias", (D,), 0.02, 0.0))
    for l in range(cfg.layers):
        q = f"model.layers.{l}."
        s.append((q + "input_layernorm.weight", (D,), 0.05, 1.0))
        for nm, n in (("q_proj", cfg.heads * hd), ("k_proj", cfg.kv_heads * hd), ("v_proj", cfg.kv_heads * hd)):
            s.append((q + f"self_attn.{nm}.weight", (n, D), 0.02, 0.0))
            s.append((q + f"self_attn.{nm}.bias", (n,), 0.02, 0.0))
        s.append((q + "self_attn.o_proj.weight", (D, cfg.heads * hd), 0.02, 0.0))
        s.append((q + "post_attention_layernorm.weight", (D,), 0.05, 1.0))
        s.append((q + "mlp.gate_proj.weight", (I, D), 0.02, 0.0))
        s.append((q + "mlp.up_proj.weight", (I, D), 0.02, 0.0))
        s.append((q + "mlp.down_proj.weight", (D, I), 0.02, 0.0))
        if cfg.lora_rank > 0:
            for nm, n_out, n_in in (("self_attn.q_proj", cfg.heads * hd, D), ("self_attn.k_proj", cfg.kv_heads * hd, D),
                                    ("self_attn.v_proj", cfg.kv_heads * hd, D), ("self_attn.o_proj", D, cfg.heads * hd),
                                    ("mlp.gate_proj", I, D), ("mlp.up_proj", I, D), ("mlp.down_proj", D, I)):
                s += lora_specs(q + nm, n_out, n_in, cfg.lora_rank)
    s.append(("model.norm.weight", (D,), 0.05, 1.0))
    if cfg.add_cross_attention:
        # W_q / W_k exist in the checkpoint but cannot influence the reward: every un-masked K row is the same
        # pad-token embedding, so the softmax is uniform whatever the scores are (rw_model:387-395).
        s.append(("W_q.weight", (D, D), 0.02, 0.0))
        s.append(("W_k.weight", (D, D), 0.02, 0.0))
        s.append(("W_v.weight", (D, D), 0.02, 0.0))
        s.append(("ca_layernorm.weight", (D,), 0.05, 1.0))
    s.append(("value_head.weight", (cfg.value_head_dim, D), 1.0 / math.sqrt(D), 0.0))
    return s


def qwen_make_weights(cfg: QwenConfig, seed: int, profile: int = 0) -> Dict[str, np.ndarray]:
    return {n: gen_tensor(seed, n, sh, std, off, profile=profile) for n, sh, std, off in qwen_weight_specs(cfg)}


def qwen_synth_batch(cfg: QwenConfig, seed: int, caption_lens: List[int], grids, with_pixels: bool = True):
    """inputs_batch of the qwen branch (what Qwen2_5_VLProcessor hands over): input_ids with the image slot
    already expanded to t*h*w/4 <|image_pad|> tokens, attention_mask (left padded with the pad token),
    pixel_values [sum t*h*w, 1176] fp32 (patches in merge-block order) and image_grid_thw [B, 3] (one image
    per row).  `grids` = (h, w) in PATCHES per row (even numbers)."""
    batch = len(caption_lens)
    if isinstance(grids[0], int):
        grids = [tuple(grids)] * batch
    assert len(grids) == batch
    v = cfg.vision
    thw = [(1, int(h), int(w)) for h, w in grids]
    special = {cfg.image_token_id, cfg.pad_token_id, QWEN_CA_TOKEN_ID}
    lo, hi = 3, min(cfg.vocab_size, min(special)) - 4
    rows = []
    for b, n in enumerate(caption_lens):
        t = tensor_seed(seed, f"caption.{b}")
        cap = np.array([lo + splitmix64_scalar((t + i) & MASK64) % (hi - lo) for i in range(n)], dtype=np.int64)
        nimg = thw[b][0] * thw[b][1] * thw[b][2] // v.merge_unit
        rows.append(np.concatenate([np.array([hi + 1, 5, hi + 2], dtype=np.int64),            # <|im_start|> user <|vision_start|>
                                    np.full(nimg, cfg.image_token_id, dtype=np.int64),
                                    np.array([hi + 3], dtype=np.int64), cap,                  # <|vision_end|> caption
                                    np.array([hi + 1], dtype=np.int64)]))
    S = max(len(r) for r in rows)
    ids = np.full((batch, S), cfg.pad_token_id, dtype=np.int64)
    mask = np.zeros((batch, S), dtype=np.int64)
    for b, row in enumerate(rows):
        ids[b, S - len(row):] = row
        mask[b, S - len(row):] = 1
    pix = None
    if with_pixels:
        pix = np.concatenate([synth_pixels(seed, f"pixel_values.{b}", (t * h * w, v.patch_dim))
                              for b, (t, h, w) in enumerate(thw)], axis=0)
    return dict(input_ids=ids, attention_mask=mask, pixel_values=pix,
                image_grid_thw=np.array(thw, dtype=np.int64))
