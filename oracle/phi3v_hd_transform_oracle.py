"""CPU oracle of the Phi-3.5-V image hand-over: uint8 RGB image -> pixel_values [num_crops+1, 3, 336, 336] fp32 and
image_sizes (h, w).  TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg); the product
path (llava_reward_amd.preprocess -> lr_hd_transform, HIP) never imports it.

Follows the reference's processor:
  * llava_reward/models/base_mllm/phi3_v/processing_phi3_v.py:62-72   padding_336 (white rows above / below)
  * :85-107   HD_transform (transpose portrait images, scale search over hd_num, resize, pad, transpose back)
  * :262-288  ToTensor + Normalize, bicubic 336x336 global view, crop tiling (global first, then local crops row-major),
              zero padding to num_crops + 1 crops, image_sizes = padded (h, w), num_img_tokens
The two third-party primitives on that path are absent from /root/reference and restated here from their published
algorithms:
  * torchvision.transforms.functional.resize on a PIL image (torchvision is NOT installed in this image) = PIL
    Image.resize((w, h), BILINEAR): Pillow's two-pass fixed-point resampler (src/libImaging/Resample.c: precompute_coeffs,
    normalize_coeffs_8bpc with PRECISION_BITS = 22, horizontal pass then vertical pass, uint8 rounding in between);
  * torch.nn.functional.interpolate(mode='bicubic', align_corners=False): cubic convolution with A = -0.75, source index
    scale * (o + 0.5) - 0.5, border taps clamped (ATen UpSampleKernel / UpSample.h).
Pinning: the resampler restatement is checked bit-for-bit against Pillow itself (tests/test_preprocess_oracle.py, Pillow
12.2 in this image and on the GPU box) and the bicubic restatement against torch on CPU (1e-6); the glue (scale search,
padding, transposition, tiling, zero crops, token count) is pinned against the reference's own Phi3VImageProcessor.preprocess,
imported in the build container with a PIL-backed stand-in for the five torchvision entry points it uses
(tests/golden/make_preprocess_goldens.py reference_processor: every pre_*.json digest is that function's output; what is pinned
under it is Pillow's resampler, which torchvision calls for PIL inputs, not torchvision's tensor path).
"""
from __future__ import annotations

import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)          # preprocessor_config.json:7-17 (OpenAI CLIP statistics)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def _bilinear_filter(x: float) -> float:
    if x < 0.0:
        x = -x
    return 1.0 - x if x < 1.0 else 0.0


def _bicubic_filter(x: float) -> float:
    a = -0.5                                                   # Pillow's BICUBIC (Keys, a = -0.5); torch's is a = -0.75
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


FILTERS = {"bilinear": (_bilinear_filter, 1.0), "bicubic": (_bicubic_filter, 2.0)}


def bilinear_coeffs(in_size: int, out_size: int):
    return resample_coeffs(in_size, out_size, "bilinear")


def resample_coeffs(in_size: int, out_size: int, filt: str = "bilinear"):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for one of Pillow's filters over the box [0, in_size).
    Returns (bounds [out, 2] = (first tap, tap count), kk [out, ksize] int32)."""
    fn, fsupport = FILTERS[filt]
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = fsupport * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = np.zeros(ksize, dtype=np.float64)
        ww = 0.0
        for x in range(xmax):
            v = fn((x + xmin - center + 0.5) * ss)
            w[x] = v
            ww += v
        for x in range(xmax):
            if ww != 0.0:
                w[x] /= ww
        for x in range(ksize):
            kk[xx, x] = int(-0.5 + w[x] * (1 << PRECISION_BITS)) if w[x] < 0 else int(0.5 + w[x] * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _resample_axis(img: np.ndarray, out_size: int, axis: int, filt: str = "bilinear") -> np.ndarray:
    """One pass of the 8-bit resampler along `axis` (0 = rows / vertical, 1 = columns / horizontal) of an [h, w, c] image."""
    in_size = img.shape[axis]
    bounds, kk = resample_coeffs(in_size, out_size, filt)
    src = np.moveaxis(img, axis, 0).astype(np.int64)              # [in, other, c]
    out = np.empty((out_size,) + src.shape[1:], dtype=np.uint8)
    for xx in range(out_size):
        x0, n = int(bounds[xx, 0]), int(bounds[xx, 1])
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), dtype=np.int64)
        for t in range(n):
            acc += src[x0 + t] * int(kk[xx, t])
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize_u8(img: np.ndarray, new_h: int, new_w: int, filt: str = "bilinear") -> np.ndarray:
    """PIL Image.resize((new_w, new_h), filt) on an RGB uint8 [h, w, 3] image: horizontal pass, then vertical pass,
    each skipped when that size does not change (Resample.c ImagingResample)."""
    h, w, _ = img.shape
    if new_w != w:
        img = _resample_axis(img, new_w, 1, filt)
    if new_h != h:
        img = _resample_axis(img, new_h, 0, filt)
    return img


def resize_bilinear_u8(img: np.ndarray, new_h: int, new_w: int) -> np.ndarray:
    return resize_u8(img, new_h, new_w, "bilinear")


def hd_geometry(width: int, height: int, hd_num: int):
    """Scale search of HD_transform (:85-101) on the landscape orientation.
    Returns (transposed, new_w, new_h, top_pad, padded_h) in the orientation the resize runs in."""
    trans = width < height
    if trans:
        width, height = height, width
    ratio = width / height
    scale = 1
    while scale * np.ceil(scale / ratio) <= hd_num:
        scale += 1
    scale -= 1
    new_w = int(scale * 336)
    new_h = int(new_w / ratio)
    tar = int(np.ceil(new_h / 336) * 336)
    top = int((tar - new_h) / 2)
    return trans, new_w, new_h, top, tar


def hd_transform(img: np.ndarray, hd_num: int = 16) -> np.ndarray:
    """HD_transform + padding_336 on an RGB uint8 [h, w, 3] array; returns the padded uint8 image [H, W, 3]."""
    h, w, _ = img.shape
    trans, new_w, new_h, top, tar = hd_geometry(w, h, hd_num)
    if trans:
        img = img.transpose(1, 0, 2)                              # Image.TRANSPOSE
    img = resize_bilinear_u8(np.ascontiguousarray(img), new_h, new_w)
    out = np.full((tar, new_w, 3), 255, dtype=np.uint8)
    out[top:top + new_h] = img
    if trans:
        out = out.transpose(1, 0, 2)
    return np.ascontiguousarray(out)


def normalize(img_u8: np.ndarray) -> np.ndarray:
    """ToTensor (u8 -> f32 / 255, CHW) then Normalize ((x - mean) / std), every step in fp32."""
    x = img_u8.astype(np.float32).transpose(2, 0, 1) / np.float32(255)
    mean = np.array(CLIP_MEAN, dtype=np.float32)[:, None, None]
    std = np.array(CLIP_STD, dtype=np.float32)[:, None, None]
    return ((x - mean) / std).astype(np.float32)


def _cubic_weights(t: np.float32):
    A = np.float32(-0.75)
    one = np.float32(1)

    def c1(x):
        return ((A + np.float32(2)) * x - (A + np.float32(3))) * x * x + one

    def c2(x):
        return ((A * x - np.float32(5) * A) * x + np.float32(8) * A) * x - np.float32(4) * A

    return c2(t + one), c1(t), c1(one - t), c2(np.float32(2) - t)


def bicubic_resize_f32(x: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """F.interpolate(x[None], size=(out_h, out_w), mode='bicubic') on a [C, H, W] fp32 array."""
    C, H, W = x.shape

    def taps(in_size, out_size):
        scale = np.float32(in_size) / np.float32(out_size)
        idx = np.zeros((out_size, 4), dtype=np.int64)
        wts = np.zeros((out_size, 4), dtype=np.float32)
        for o in range(out_size):
            real = scale * (np.float32(o) + np.float32(0.5)) - np.float32(0.5)
            i0 = math.floor(real)
            t = np.float32(real - np.float32(i0))
            wts[o] = _cubic_weights(t)
            idx[o] = [min(max(i0 - 1 + j, 0), in_size - 1) for j in range(4)]
        return idx, wts

    iy, wy = taps(H, out_h)
    ix, wx = taps(W, out_w)
    out = np.zeros((C, out_h, out_w), dtype=np.float32)
    for i in range(4):
        rows = x[:, iy[:, i], :]                                   # [C, out_h, W]
        inner = np.zeros((C, out_h, out_w), dtype=np.float32)
        for j in range(4):
            inner += rows[:, :, ix[:, j]] * wx[None, None, :, j]
        out += inner * wy[None, :, i, None]
    return out


def num_img_tokens(h: int, w: int) -> int:
    return int(((h // 336) * (w // 336) + 1) * 144 + 1 + (h // 336 + 1) * 12)     # :269


def preprocess(img: np.ndarray, num_crops: int = 16):
    """Phi3VImageProcessor.preprocess for one RGB uint8 image (:262-288).
    Returns (pixel_values [num_crops+1, 3, 336, 336] f32, (h, w) of the padded image, num_img_tokens)."""
    hd = normalize(hd_transform(img, num_crops))                  # [3, H, W]
    _, H, W = hd.shape
    glob = bicubic_resize_f32(hd, 336, 336)
    local = hd.reshape(3, H // 336, 336, W // 336, 336).transpose(1, 3, 0, 2, 4).reshape(-1, 3, 336, 336)
    out = np.zeros((num_crops + 1, 3, 336, 336), dtype=np.float32)
    out[0] = glob
    out[1:1 + local.shape[0]] = local
    return out, (H, W), num_img_tokens(H, W)
